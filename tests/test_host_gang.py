"""Gang scheduling inputs on the CPU (mirror-only handle): the task-groups annotation is validated like
GetTaskGroupsFromAnnotation / validateTaskGroupResources (/root/reference/pkg/cache/utils.go:33-121) and expanded into
placeholder asks like newPlaceholder (/root/reference/pkg/cache/placeholder.go:40-157). Annotations and expectations are
those of /root/reference/pkg/cache/utils_test.go:43-259 and placeholder_test.go."""
import importlib
import itertools
import json
from fractions import Fraction

import pytest

pkg = importlib.import_module("yunikorn-k8shim_amd")
MAX = 2**63 - 1


@pytest.fixture()
def mirror():
    m = pkg.GpuPredicateManager(device=-1)
    yield m
    m.close()


GOOD = """
[
  {"name": "test-group-1", "minMember": 10, "minResource": {"cpu": 1, "memory": "2Gi"},
   "nodeSelector": {"test": "testnode", "locate": "west"},
   "tolerations": [{"key": "key", "operator": "Equal", "value": "value", "effect": "NoSchedule"}]},
  {"name": "test-group-2", "minMember": 5, "minResource": {"cpu": 2, "memory": "4Gi"}}
]"""
BAD = {  # utils_test.go:86-146
    "malformed json / wrong types": '[{"name": "test-group-err-1", "minMember": "ERR", "minResource": {"cpu": "ERR", "memory": "ERR"},}]',
    "without name": '[{"minMember": 3, "minResource": {"cpu": 2, "memory": "1Gi"}}]',
    "without minMember": '[{"name": "test-group-err-3", "minResource": {"cpu": 2, "memory": "1Gi"}}]',
    "without minResource": '[{"name": "test-group-err-4", "minMember": 3}]',
    "negative minMember without minResource": '[{"name": "test-group-err-5", "minMember": -100}]',
    "negative minMember with minResource": '[{"name": "test-group-err-6", "minMember": -100, "minResource": {"cpu": 2, "memory": "1Gi"}}]',
}
REJECTED = {  # utils_test.go:231-244
    "negative cpu": '[{"name":"g","minMember":1,"minResource":{"cpu":"-1"}}]',
    "cpu MilliValue overflow": '[{"name":"g","minMember":1,"minResource":{"cpu":"9223372036854776"}}]',
    "memory Value overflow": '[{"name":"g","minMember":1,"minResource":{"memory":"9223372036854775808"}}]',
    "minMember times minResource overflow": '[{"name":"g","minMember":2000000000,"minResource":{"cpu":"5000000"}}]',
    "aggregate overflow across taskGroups": '[{"name":"a","minMember":1,"minResource":{"memory":"5E"}},{"name":"b","minMember":1,"minResource":{"memory":"5E"}}]',
    "cpu and vcore canonical aggregate overflow": '[{"name":"a","minMember":1,"minResource":{"cpu":"5P"}},{"name":"b","minMember":1,"minResource":{"vcore":"5E"}}]',
    "cpu and vcore collide within a group": '[{"name":"g","minMember":1,"minResource":{"cpu":"1","vcore":"1"}}]',
    "explicit pods collides with implicit pods": '[{"name":"a","minMember":2147483647,"minResource":{}},{"name":"b","minMember":1,"minResource":{"pods":"9223372036854775807"}}]',
}


def test_accepts_the_reference_annotations(mirror):
    assert mirror.validate_task_groups(GOOD) == 2
    assert mirror.validate_task_groups('[{"name": "test-group-3", "minMember": 3, "minResource": {"cpu": 2, "memory": "1Gi"}}]') == 1
    assert mirror.validate_task_groups('[{"name":"g","minMember":2,"minResource":{"cpu":"1","memory":"1Gi"}}]') == 1  # :251-255


@pytest.mark.parametrize("name", list(BAD) + list(REJECTED))
def test_rejects_what_the_reference_rejects(mirror, name):
    with pytest.raises(RuntimeError):
        mirror.validate_task_groups((BAD | REJECTED)[name])
    with pytest.raises(RuntimeError):
        mirror.add_task_groups("app", "root.q", "ns", (BAD | REJECTED)[name])
    assert mirror.num_pods == 0


SUFFIX = {"": 1, "m": Fraction(1, 1000), "Gi": 2**30, "E": 10**18, "P": 10**15}


def exact(q):
    for suf in ("Gi", "m", "E", "P", ""):
        if q.endswith(suf) and (suf or q[-1].isdigit()):
            return Fraction(q[:len(q) - len(suf)] if suf else q) * SUFFIX[suf]
    raise ValueError(q)


def ceil(fr):
    return -((-fr.numerator) // fr.denominator)


def reference_verdict(groups):
    """validateTaskGroupResources with exact arithmetic: (accepted, exact placeholder ask per canonical key)."""
    totals = {"pods": 0}
    for g in groups:
        members = g["minMember"]
        seen = {"pods"}
        if totals["pods"] > MAX - members:
            return False, None
        totals["pods"] += members
        for name, q in g["minResource"].items():
            val = exact(q)
            cpu = name == "cpu"
            canonical = "vcore" if cpu else name
            if val < 0 or canonical in seen:
                return False, None
            seen.add(canonical)
            if val > MAX // (members * (1000 if cpu else 1)):
                return False, None
            contribution = members * ceil(val * (1000 if cpu else 1))
            if totals.get(canonical, 0) > MAX - contribution:
                return False, None
            totals[canonical] = totals.get(canonical, 0) + contribution
    return True, totals


def test_validator_matches_exact_arithmetic_on_the_reference_grid(mirror):
    """The grid of TestValidatedTaskGroupsNeverProduceNegativePlaceholderAsk (utils_test.go:262-330): every combination is
    accepted or rejected exactly as integer-exact arithmetic says, and an accepted ask always fits int64."""
    strs = ["0", "1", "500m", "1.5", "2", "1Gi", "5E", "5P", "9223372036854775807", "9223372036854775808", "18446744073709551616"]
    keys = ["cpu", "vcore", "memory", "pods", "nvidia.com/gpu"]
    accepted = 0
    for k1, k2, s, m in itertools.product(keys, keys, strs, [1, 2, 2000000000, 2**31 - 1]):
        groups = [{"name": "a", "minMember": m, "minResource": {k1: s}}, {"name": "b", "minMember": 1, "minResource": {k2: s}}]
        want, totals = reference_verdict(groups)
        try:
            mirror.validate_task_groups(groups)
            got = True
        except RuntimeError:
            got = False
        assert got == want, groups
        if got:
            accepted += 1
            assert all(0 <= v <= MAX for v in totals.values())
    assert accepted > 0


def test_placeholders_are_built_like_new_placeholder(mirror):
    gpu, huge = "nvidia.com/gpu", "hugepages-1Gi"
    groups = [
        {"name": "test-group-1", "minMember": 10,
         "minResource": {"cpu": "500m", "memory": "1024M", gpu: "2", huge: "2Gi", "ephemeral-storage": "1Gi"},  # placeholder_test.go: 5 requests
         "labels": {"labelKey0": "labelKeyValue0", "yunikorn.apache.org/queue": "overridden-by-the-app"},
         "nodeSelector": {"nodeType": "test"},
         "tolerations": [{"key": "key1", "operator": "Equal", "value": "value1", "effect": "NoSchedule"}],
         "affinity": {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": [
             {"matchExpressions": [{"key": "zone", "operator": "In", "values": ["a", "b"]}]}]}}},
         "topologySpreadConstraints": [{"maxSkew": 1, "topologyKey": "topology.kubernetes.io/zone", "whenUnsatisfiable": "DoNotSchedule",
                                        "labelSelector": {"matchLabels": {"labelKey0": "labelKeyValue0", "labelKey1": "labelKeyValue1"}}}]},
        {"name": "a-task-group-name-longer-than-twenty-characters", "minMember": 3, "minResource": {"cpu": "1", "": "7"}},
    ]
    app = "application-id-that-is-longer-than-28-characters"
    assert mirror.add_task_groups(app, "root.default", "test-ns", groups) == 13
    assert mirror.num_pods == 13
    pods = json.loads(mirror.dump_snapshot())["pods"]
    first, other = pods[:10], pods[10:]
    for p in first:
        assert p["metadata"]["name"].startswith(f"tg-{app[:28]}-test-group-1-") and len(p["metadata"]["name"]) <= 63
        assert p["metadata"]["namespace"] == "test-ns"
        assert p["metadata"]["labels"] == {"labelKey0": "labelKeyValue0", "yunikorn.apache.org/app-id": app,
                                           "yunikorn.apache.org/queue": "root.default"}
        spec = p["spec"]
        assert spec["nodeSelector"] == {"nodeType": "test"}
        assert spec["tolerations"][0]["key"] == "key1" and spec["tolerations"][0]["effect"] == "NoSchedule"
        tsc = spec["topologySpreadConstraints"][0]
        assert (tsc["maxSkew"], tsc["topologyKey"], tsc["whenUnsatisfiable"]) == (1, "topology.kubernetes.io/zone", "DoNotSchedule")
        assert tsc["labelSelector"]["matchLabels"] == {"labelKey0": "labelKeyValue0", "labelKey1": "labelKeyValue1"}
        assert spec["affinity"]["nodeAffinity"]["requiredDuringSchedulingIgnoredDuringExecution"]["nodeSelectorTerms"][0][
            "matchExpressions"][0]["values"] == ["a", "b"]
    assert len({p["metadata"]["name"] for p in pods}) == 13
    assert all(p["metadata"]["name"].startswith(f"tg-{app[:28]}-a-task-group-name-lo-") for p in other)
    # requests = minResource (five entries for the first group; the empty resource name of the second is dropped)
    assert mirror.pod_request(0) == {"cpu": 500, "memory": 1024000000, gpu: 2, huge: 2 << 30, "ephemeral-storage": 1 << 30}
    assert mirror.pod_request(12) == {"cpu": 1000}
    # every member of a group shares one interned template: two templates for thirteen asks
    assert mirror.stats()["templates"] == 2
