"""Small seeded random clusters (Kubernetes-JSON snapshots) that hit the edge cases of the predicate path.
Independent of the product's own KWOK generator: pure Python, used only by tests."""
import random

ZONES = [f"z{i}" for i in range(4)]
TEAMS = ["a", "b", "c"]
EFFECTS = ["NoSchedule", "NoExecute", "PreferNoSchedule"]
CPU = ["0", "1m", "100m", "250m", "500m", "1", "2", "3500m", "8"]
MEM = ["0", "1", "1M", "128Mi", "1Gi", "4Gi", "1e9", "1500M"]


def rand_port(rng):
    """A container port; hostPort 0 = no host port (ignored by NodePorts)."""
    p = {"hostPort": rng.choice([0, 80, 80, 443, 8080, 9090]), "containerPort": 8000}
    if rng.random() < 0.6:
        p["protocol"] = rng.choice(["TCP", "UDP", "SCTP"])
    if rng.random() < 0.5:
        p["hostIP"] = rng.choice(["0.0.0.0", "127.0.0.1", "10.0.0.1", ""])
    return p


def rand_node(rng, i, scalars=True, interpod=False):
    name = f"node-{i}" if not (i == 0 and rng.random() < 0.3) else ""
    labels = {}
    if rng.random() < 0.9:
        labels["zone"] = rng.choice(ZONES)
    if rng.random() < 0.5:
        labels["kernel-version"] = rng.choice(["0204", "0206", "0510", "abc", "-3", "+7"])
    if rng.random() < 0.3:
        labels["gpu"] = rng.choice(["a100", "mi355x", ""])
    if rng.random() < 0.5:
        labels["example.com/tier"] = rng.choice(["gold", "silver"])
    if name:
        labels["kubernetes.io/hostname"] = name
    taints = []
    for _ in range(rng.choice([0, 0, 1, 1, 2, 3])):
        taints.append({"key": rng.choice(["dedicated", "kwok.x-k8s.io/node", "maint"]), "value": rng.choice(TEAMS + ["", "fake"]),
                       "effect": rng.choice(EFFECTS)})
    alloc = {"cpu": rng.choice(["4", "8", "16", "500m", "0"]), "memory": rng.choice(["8Gi", "16Gi", "1Gi", "0"]),
             "pods": rng.choice(["110", "3", "1", "0"])}
    if rng.random() < 0.5:
        alloc["ephemeral-storage"] = rng.choice(["10Gi", "1Gi"])
    if scalars and rng.random() < 0.4:
        alloc["example.com/gpu"] = rng.choice(["0", "1", "4", "8"])
    if scalars and rng.random() < 0.2:
        alloc["hugepages-2Mi"] = rng.choice(["1Gi", "0"])
    pods = []
    for j in range(rng.choice([0, 0, 1, 2, 3, 5])):
        req = {}
        if rng.random() < 0.8:
            req["cpu"] = rng.choice(CPU)
        if rng.random() < 0.8:
            req["memory"] = rng.choice(MEM)
        if scalars and rng.random() < 0.2:
            req["example.com/gpu"] = rng.choice(["1", "2"])
        meta = {"name": f"n{i}-p{j}", "uid": f"n{i}-p{j}", "namespace": rng.choice(["default", "default", "default", "other"]),
                "labels": {"app": rng.choice(TEAMS)}}
        if rng.random() < 0.05:
            meta["deletionTimestamp"] = "2026-01-01T00:00:00Z"  # terminating pods are not counted by PodTopologySpread
        cont = {"name": "c", "resources": {"requests": req}}
        if rng.random() < 0.15:
            cont["ports"] = [rand_port(rng) for _ in range(rng.choice([1, 1, 2]))]
        entry = {"metadata": meta, "spec": {"containers": [cont]}}
        if interpod and rng.random() < 0.2:
            entry["spec"]["affinity"] = rand_pod_affinity(rng)
        if rng.random() < 0.2:
            entry["replicas"] = rng.choice([2, 3])
        pods.append(entry)
    node = {"metadata": {"name": name, "labels": labels}, "spec": {"taints": taints, "unschedulable": rng.random() < 0.15},
            "status": {"allocatable": alloc}, "pods": pods}
    return node


def rand_requirement(rng, n_nodes):
    op = rng.choice(["In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt", "In", "NotIn", "Bogus"])
    key = rng.choice(["zone", "kernel-version", "gpu", "example.com/tier", "kubernetes.io/hostname", "bad key!", "missing"])
    r = {"key": key, "operator": op}
    if op in ("In", "NotIn"):
        pool = ZONES + ["gold", "silver", "a100", "", f"node-{rng.randrange(max(n_nodes, 1))}", "invalid value: ___@#$%^"]
        k = rng.choice([0, 1, 1, 2, 3])
        vals = [rng.choice(pool) for _ in range(k)]
        if vals or rng.random() < 0.5:
            r["values"] = vals
    elif op in ("Gt", "Lt"):
        r["values"] = rng.choice([["0205"], ["0"], ["-5"], ["x"], ["1", "2"], []])
    elif rng.random() < 0.1:
        r["values"] = ["unexpected"]
    return r


def rand_field(rng, n_nodes):
    op = rng.choice(["In", "In", "NotIn", "Exists"])
    key = rng.choice(["metadata.name", "metadata.name", "metadata.name", "metadata.namespace"])
    vals = [rng.choice([f"node-{rng.randrange(max(n_nodes, 1))}", "nope", ""]) for _ in range(rng.choice([1, 1, 1, 2, 0]))]
    return {"key": key, "operator": op, "values": vals}


def rand_pod_terms(rng):
    """1-2 required pod (anti)affinity terms over the app / tier labels and a few topology keys."""
    terms = []
    for _ in range(rng.choice([1, 1, 2])):
        k = rng.random()
        if k < 0.5:
            selector = {"matchLabels": {"app": rng.choice(TEAMS)}}
        elif k < 0.8:
            selector = {"matchExpressions": [{"key": "app", "operator": rng.choice(["In", "NotIn"]), "values": rng.sample(TEAMS, rng.choice([1, 2]))}]}
        elif k < 0.9:
            selector = {"matchExpressions": [{"key": "app", "operator": "Exists"}]}
        else:
            selector = {}
        t = {"labelSelector": selector, "topologyKey": rng.choice(["zone", "zone", "kubernetes.io/hostname", "example.com/tier", "missing-key"])}
        if rng.random() < 0.25:
            t["namespaces"] = rng.sample(["default", "other", "third"], rng.choice([1, 2]))
        if rng.random() < 0.05:
            t.pop("labelSelector")
        terms.append(t)
    return terms


def rand_pod_affinity(rng):
    a = {}
    if rng.random() < 0.5:
        a["podAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": rand_pod_terms(rng)}
    if rng.random() < 0.6 or not a:
        a["podAntiAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": rand_pod_terms(rng)}
    return a


def rand_spread(rng):
    """0-2 topologySpreadConstraints with distinct topology keys (API validation forbids duplicate key+whenUnsatisfiable)."""
    out = []
    keys = ["zone", "kubernetes.io/hostname", "example.com/tier"]
    rng.shuffle(keys)
    for key in keys[:rng.choice([1, 1, 2])]:
        c = {"maxSkew": rng.choice([1, 1, 2, 3]), "topologyKey": key,
             "whenUnsatisfiable": rng.choice(["DoNotSchedule", "DoNotSchedule", "DoNotSchedule", "ScheduleAnyway"])}
        k = rng.random()
        if k < 0.1:
            pass  # nil selector: matches nothing
        elif k < 0.2:
            c["labelSelector"] = {}
        elif k < 0.75:
            c["labelSelector"] = {"matchLabels": {"app": rng.choice(TEAMS)}}
        else:
            c["labelSelector"] = {"matchExpressions": [{"key": "app", "operator": rng.choice(["In", "NotIn", "Exists"]),
                                                        **({"values": rng.sample(TEAMS, 2)} if rng.random() < 0.7 else {})}]}
            if c["labelSelector"]["matchExpressions"][0]["operator"] == "Exists":
                c["labelSelector"]["matchExpressions"][0].pop("values", None)
            elif "values" not in c["labelSelector"]["matchExpressions"][0]:
                c["labelSelector"]["matchExpressions"][0]["values"] = [rng.choice(TEAMS)]
        if rng.random() < 0.25:
            c["minDomains"] = rng.choice([1, 2, 5, 50])
        if rng.random() < 0.3:
            c["nodeAffinityPolicy"] = rng.choice(["Honor", "Ignore"])
        if rng.random() < 0.3:
            c["nodeTaintsPolicy"] = rng.choice(["Honor", "Ignore"])
        out.append(c)
    return out


def rand_pod(rng, i, n_nodes, scalars=True, spread=False, interpod=False):
    spec = {}
    req = {}
    if rng.random() < 0.8:
        req["cpu"] = rng.choice(CPU)
    if rng.random() < 0.8:
        req["memory"] = rng.choice(MEM)
    if rng.random() < 0.2:
        req["ephemeral-storage"] = rng.choice(["1Gi", "20Gi"])
    if scalars and rng.random() < 0.3:
        req["example.com/gpu"] = rng.choice(["0", "1", "2", "8"])
    if scalars and rng.random() < 0.1:
        req["hugepages-2Mi"] = "512Mi"
    containers = [{"name": "main", "resources": {"requests": req}}]
    if rng.random() < 0.25:
        containers[0]["ports"] = [rand_port(rng) for _ in range(rng.choice([1, 2, 3]))]
    if rng.random() < 0.3:
        containers.append({"name": "side", "resources": {"requests": {"cpu": rng.choice(CPU)}}})
    spec["containers"] = containers
    if rng.random() < 0.25:
        ics = []
        for k in range(rng.choice([1, 2, 3])):
            ic = {"name": f"ic{k}", "resources": {"requests": {"cpu": rng.choice(CPU), "memory": rng.choice(MEM)}}}
            if rng.random() < 0.4:
                ic["restartPolicy"] = "Always"
            if rng.random() < 0.2:
                ic["ports"] = [rand_port(rng)]
            ics.append(ic)
        spec["initContainers"] = ics
    if rng.random() < 0.15:
        spec["overhead"] = {"cpu": "100m", "memory": "64Mi"}
    if rng.random() < 0.1:
        spec["resources"] = {"requests": {"cpu": rng.choice(CPU), "memory": rng.choice(MEM), "example.com/gpu": "64"}}
    tols = []
    for _ in range(rng.choice([0, 0, 1, 2, 3])):
        kind = rng.random()
        if kind < 0.15:
            tols.append({"operator": "Exists"})
        elif kind < 0.3:
            tols.append({"key": "node.kubernetes.io/unschedulable", "operator": "Exists", "effect": rng.choice(["NoSchedule", ""])})
        else:
            t = {"key": rng.choice(["dedicated", "kwok.x-k8s.io/node", "maint", ""]), "operator": rng.choice(["Exists", "Equal", "", "Lt"]),
                 "value": rng.choice(TEAMS + ["", "fake"]), "effect": rng.choice(EFFECTS + [""])}
            tols.append(t)
    if tols:
        spec["tolerations"] = tols
    r = rng.random()
    if r < 0.25:
        spec["nodeSelector"] = rng.choice([{}, {"zone": rng.choice(ZONES)}, {"zone": rng.choice(ZONES), "example.com/tier": "gold"},
                                          {"bad key!": "x"}, {"gpu": ""}])
    if 0.15 < r < 0.7:
        kind = rng.random()
        if kind < 0.08:
            terms = None
        elif kind < 0.16:
            terms = []
        else:
            terms = []
            for _ in range(rng.choice([1, 1, 2, 3])):
                term = {}
                if rng.random() < 0.85:
                    term["matchExpressions"] = [rand_requirement(rng, n_nodes) for _ in range(rng.choice([0, 1, 1, 2]))]
                if rng.random() < 0.35:
                    term["matchFields"] = [rand_field(rng, n_nodes) for _ in range(rng.choice([1, 1, 2]))]
                terms.append(term)
        spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": {"nodeSelectorTerms": terms}}}
    elif r > 0.95:
        spec["affinity"] = {"nodeAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": None}}
    if rng.random() < 0.1:
        spec["nodeName"] = rng.choice([f"node-{rng.randrange(max(n_nodes, 1))}", "ghost"])
    if spread and rng.random() < 0.5:
        spec["topologySpreadConstraints"] = rand_spread(rng)
    if interpod and rng.random() < 0.5:
        spec.setdefault("affinity", {}).update(rand_pod_affinity(rng))
    ns = rng.choice(["default", "default", "other"]) if interpod else "default"
    return {"metadata": {"name": f"pod-{i}", "uid": f"pod-{i}", "namespace": ns, "labels": {"app": rng.choice(TEAMS)}}, "spec": spec}


def random_snapshot(seed, n_nodes, n_pods, scalars=True, spread=False, interpod=False):
    rng = random.Random(seed)
    nodes = [rand_node(rng, i, scalars, interpod) for i in range(n_nodes)]
    pods = [rand_pod(rng, i, n_nodes, scalars, spread, interpod) for i in range(n_pods)]
    # a few exact duplicates so that classes have several members
    for i in range(min(n_pods // 4, 16)):
        src = pods[rng.randrange(len(pods))]
        dup = {"metadata": dict(src["metadata"], name=f"dup-{i}", uid=f"dup-{i}"), "spec": src["spec"]}
        pods.append(dup)
    return {"nodes": nodes, "pods": pods}
